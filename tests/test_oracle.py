"""CPU tests that pin the oracle (oracle/pyfft_oracle.py, oracle/fft_oracle.c) against the golden
fixtures and the reference's own known-answer cases.  No GPU."""
import json
import os

import numpy
import pytest

import pyfft_oracle as oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EPS = {"complex64": 1.1e-6, "complex128": 1e-11}   # test/test_errors.py:20-23


@pytest.fixture(scope="module")
def vectors():
    z = numpy.load(os.path.join(GOLDEN, "fft_vectors.npz"))
    index = json.loads(bytes(z["index_json"]).decode())
    return z, index


@pytest.fixture(scope="module")
def decomps():
    with open(os.path.join(GOLDEN, "ref_decompositions.json")) as f:
        return json.load(f)


def test_radix_decompositions_match_reference(decomps):
    """The oracle's restated getGlobalRadixInfo / getRadixArray equal the reference's own outputs
    (golden file produced by importing pyfft/kernel_helpers.py)."""
    for n, want in decomps["getGlobalRadixInfo"].items():
        got = oracle.get_global_radix_info(int(n))
        assert [list(g) for g in got] == want, n
    for n, want in decomps["getRadixArray_0"].items():
        assert oracle.get_radix_array(int(n), 0) == want, n
    for n, want in decomps["getRadixArray_16"].items():
        assert oracle.get_radix_array(int(n), 16) == want, n


def test_survey_chain_table():
    """SURVEY.md Appendix B rows (reference chains evaluated from plan.py:135-171)."""
    c = oracle.plan_chain((1 << 20,), numpy.complex64)
    assert [(k.R, k.M, k.S) for k in c] == [(128, 8192, 1), (128, 64, 128), (64, 1, 16384)]
    assert [k.in_place_possible for k in c] == [False, False, True]
    c = oracle.plan_chain((1024, 1024), numpy.complex64)
    assert repr(c[0]).startswith("local1024[16, 16, 4]")
    assert [(k.R, k.M, k.S) for k in c[1:]] == [(128, 8, 1024), (8, 1, 131072)]
    c = oracle.plan_chain((256, 256, 256), numpy.complex128)
    assert len(c) == 5
    c = oracle.plan_chain((2048,), numpy.complex128)      # fp64 LDS limit 1024 (plan.py:46)
    assert [k.kind for k in c] == ["global", "global"]


def test_oracle_matches_golden_vectors(vectors):
    z, index = vectors
    for ent in index:
        data = z[ent["key"] + "_in"]
        want = z[ent["key"] + "_fw"]
        shape = tuple(ent["shape"])
        got = oracle.execute(data, shape, batch=ent["batch"])
        err = oracle.difference(want, got, ent["batch"])
        assert err < EPS[ent["dtype"]], (ent, err)
        back = oracle.execute(got, shape, batch=ent["batch"], inverse=True)
        assert oracle.difference(data, back, ent["batch"]) < EPS[ent["dtype"]], ent


def test_c_oracle_matches_golden_vectors(vectors):
    import c_oracle
    if not c_oracle.available():
        pytest.skip("oracle/liboracle.so not built (run __graft_entry__.build())")
    z, index = vectors
    for ent in index:
        data = z[ent["key"] + "_in"]
        want = z[ent["key"] + "_fw"]
        _, xyz = oracle.normalize_shape(tuple(ent["shape"]))
        got = c_oracle.execute(data, xyz, batch=ent["batch"])
        assert oracle.difference(want, got, ent["batch"]) < EPS[ent["dtype"]], ent
        size = xyz[0] * xyz[1] * xyz[2]
        back = c_oracle.execute(got, xyz, batch=ent["batch"], inverse=True, divisor=float(size))
        assert oracle.difference(data, back, ent["batch"]) < EPS[ent["dtype"]], ent
        # same chain as the numpy restatement
        prec = 1 if ent["dtype"] == "complex128" else 0
        chain = oracle.plan_chain(tuple(ent["shape"]), numpy.dtype(ent["dtype"]))
        flat = []
        for k in chain:
            if k.kind == "local":
                S, cn = 1, k.n
                for R in k.radix_array:
                    flat.append((R, cn // R, S))
                    S *= R
                    cn //= R
            else:
                flat.append((k.R, k.M, k.S))
        assert c_oracle.plan(xyz[0], xyz[1], xyz[2], prec) == flat


def test_doc_known_answer():
    """doc/source/index.rst:65-99: ones((16,16)) -> 256 at [0,0], zeros elsewhere -> back to ones."""
    data = numpy.ones((16, 16), dtype=numpy.complex64)
    fw = oracle.execute(data, (16, 16))
    assert abs(fw[0, 0] - 256) < 1e-4
    fw2 = fw.copy()
    fw2[0, 0] = 0
    assert numpy.abs(fw2).max() < 1e-4
    back = oracle.execute(fw, (16, 16), inverse=True)
    assert numpy.abs(back - data).sum() / data.size < 1e-6


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128])
def test_normalize_and_scale_semantics(dtype):
    """test/test_functionality.py:53-100."""
    data = numpy.ones(16, dtype=dtype)
    for normalize in (True, False):
        fw = oracle.execute(data, (16,), normalize=normalize)
        assert numpy.abs(numpy.fft.fft(data) - fw).sum() / 16 < 1e-6
        back = oracle.execute(fw, (16,), inverse=True, normalize=normalize)
        coeff = 1 if normalize else 16
        assert numpy.abs(data * coeff - back).sum() / 16 < 1e-6
    for scale in (1.0, 10.0):
        fw = oracle.execute(data, (16,), scale=scale)
        assert numpy.abs(numpy.fft.fft(data) * scale - fw).sum() / 16 < 1e-6
        back = oracle.execute(fw, (16,), inverse=True, scale=scale)
        assert numpy.abs(data - back).sum() / 16 < 1e-6


def test_oracle_direct_vs_numpy_more_shapes():
    for shape, dt, batch in [((8192,), numpy.complex64, 2), ((1 << 14,), numpy.complex128, 1), ((256, 64), numpy.complex64, 2),
                             ((4, 1024), numpy.complex128, 1), ((2, 4, 2), numpy.complex64, 5), ((128, 16, 16), numpy.complex64, 1)]:
        data = oracle.get_test_data(shape, dt, batch, 42)
        fw = oracle.execute(data, shape, batch=batch)
        ref = oracle.numpy_fft(numpy.fft.fftn, data, batch)
        assert oracle.difference(ref, fw, batch) < EPS[numpy.dtype(dt).name]
        assert numpy.abs(fw - ref).max() <= 1e-5 * numpy.abs(ref).max()


def test_linearity_and_parseval():
    shape, batch = (2048,), 2
    a = oracle.get_test_data(shape, numpy.complex128, batch, 1)
    b = oracle.get_test_data(shape, numpy.complex128, batch, 2)
    fa, fb = oracle.execute(a, shape, batch=batch), oracle.execute(b, shape, batch=batch)
    fab = oracle.execute(2.5 * a - 1j * b, shape, batch=batch)
    assert numpy.abs(fab - (2.5 * fa - 1j * fb)).max() < 1e-9
    assert abs((numpy.abs(fa) ** 2).sum() / 2048 - (numpy.abs(a) ** 2).sum()) < 1e-6


def test_error_cases():
    """plan.py:24,48,87,89 / test_functionality.py:129-139."""
    with pytest.raises(ValueError):
        oracle.plan_chain((17,), numpy.complex64)
    with pytest.raises(ValueError):
        oracle.plan_chain((16,), numpy.int32)
    with pytest.raises(ValueError):
        oracle.plan_chain((16, 16, 16, 16), numpy.complex64)
    with pytest.raises(ValueError):
        oracle.plan_chain("16", numpy.complex64)


def test_buffer_schedule_contract():
    """plan.py:200-248: out-of-place never writes data_in; result lands in data_out (or data_in when
    in place); every non-in-place-capable kernel reads and writes different buffers."""
    for shape in [(1 << 20,), (8192,), (1024, 1024), (256, 256, 256), (16, 16), (1 << 22,), (1024, 16, 128)]:
        chain = oracle.plan_chain(shape, numpy.complex64)
        for inplace in (False, True):
            temp, sched = oracle.buffer_schedule(chain, inplace)
            assert sched[-1][1] == 1
            loc = 1 if inplace else 0
            for k, (r, w) in zip(chain, sched):
                if inplace and r == 0:
                    r = 1          # in place: data_out aliases data_in
                assert r == loc
                if not inplace:
                    assert w != 0
                if not k.in_place_possible:
                    assert r != w
                assert (2 in (r, w)) <= temp
                loc = w
