"""Pins the oracle's elementary arithmetic to the reference's own vector library: the golden
file holds outputs of Vec4d mul_add/horizontal_add/exp/log compiled from
/root/reference/vectorclass (oracle/vcl_probe.cpp, tests/make_vcl_golden.py)."""
import ctypes as C
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden", "vcl_probe.json")


def _ulps(a, b):
    a, b = np.float64(a), np.float64(b)
    return abs(int(a.view(np.int64)) - int(b.view(np.int64)))


def test_dot_product_association_is_bit_exact(oracle):
    L = oracle.lib()
    L.oracle_dot4.restype = C.c_double
    L.oracle_dot4.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
    data = json.load(open(GOLD))
    for d in data["dots"]:
        a = np.array(d["a"]); b = np.array(d["b"])
        r = L.oracle_dot4(a.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)), d["n"])
        assert r == d["r"], (d["n"], r, d["r"])  # same lanes, same unfused mul_add, same hadd tree


def test_exp_log_within_one_ulp_of_vcl(oracle):
    L = oracle.lib()
    L.oracle_exp.restype = C.c_double
    L.oracle_exp.argtypes = [C.c_double]
    L.oracle_log.restype = C.c_double
    L.oracle_log.argtypes = [C.c_double]
    data = json.load(open(GOLD))
    worst_e = max(_ulps(L.oracle_exp(x), y) for x, y in data["exp"])
    worst_l = max(_ulps(L.oracle_log(x), y) for x, y in data["log"])
    # VCL's polynomials are not correctly rounded; libm is.  <= 1 ulp apart on these ranges.
    assert worst_e <= 1 and worst_l <= 1, (worst_e, worst_l)


def test_float8_dot_product_is_bit_exact(oracle):
    """UFBoot's dotProductSIMD<float, Vec8f, 8>: the oracle's scalar restatement vs the reference's own
    Vec8f mul_add / horizontal_add (probe built from /root/reference/vectorclass)."""
    data = json.load(open(GOLD))
    assert len(data["dots8f"]) == 10
    for d in data["dots8f"]:
        x = np.array(d["x"], dtype=np.float32)
        y = np.array(d["y"], dtype=np.float32)
        assert x.size == d["n"]
        got = oracle.dot_float8(x, y)
        assert np.float32(got) == np.float32(d["r"]), (d["n"], got, d["r"])
