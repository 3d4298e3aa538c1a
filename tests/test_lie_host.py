"""lie_dev.h on the host (the same header the kernels and ba_api.hip's host code include): se3_exp against the closed forms, across the switch between the
coefficient series and sin / cos, and against scipy's matrix exponential; exp / log, mul / inv round trips.  The kernels' control steps call these on their critical
lanes (gn_tail, the tracker's leader): the header is tuned for instruction count, this pins what it computes."""
import os, subprocess, sys, tempfile
import numpy as np
import pytest
from scipy.linalg import expm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include "lie_dev.h"
int main() {
    double xi[6];
    while (scanf("%lf %lf %lf %lf %lf %lf", xi, xi + 1, xi + 2, xi + 3, xi + 4, xi + 5) == 6) {
        double T[12], Ti[12], P[12], back[6];
        ld::se3_exp(xi, T);
        ld::se3_inv(T, Ti);
        ld::se3_mul(T, Ti, P);
        ld::se3_log(T, back);
        for (int i = 0; i < 12; i++) printf("%.17g ", T[i]);
        for (int i = 0; i < 12; i++) printf("%.17g ", P[i]);
        for (int i = 0; i < 6; i++) printf("%.17g ", back[i]);
        printf("\n");
    }
}
'''


@pytest.fixture(scope="module")
def lie_bin():
    d = tempfile.mkdtemp(prefix="lie_host_")
    src, exe = os.path.join(d, "lie.cc"), os.path.join(d, "lie")
    open(src, "w").write(SRC)
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "ldso_amd", "csrc"), src, "-o", exe])
    return exe


def run(exe, xis):
    inp = "\n".join(" ".join(repr(float(v)) for v in x) for x in xis) + "\n"
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout
    rows = np.array([[float(v) for v in l.split()] for l in out.strip().splitlines()])
    return rows[:, :12].reshape(-1, 3, 4), rows[:, 12:24].reshape(-1, 3, 4), rows[:, 24:30]


def hat6(xi):
    u, w = xi[:3], xi[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = u
    return M


def test_se3_exp_matches_the_matrix_exponential_on_both_sides_of_the_series_switch(lie_bin):
    rng = np.random.default_rng(7)
    xis = []
    for scale in (1e-9, 1e-4, 1e-2, 0.2, 0.45, 0.499, 0.5, 0.501, 0.6, 1.5, 3.0):      # |omega|: the series take over below 0.5 (LD_EXP_SERIES_TH2 = 0.25)
        for _ in range(20):
            w = rng.normal(size=3); w *= scale / np.linalg.norm(w)
            xis.append(np.concatenate([rng.normal(size=3), w]))
    T, P, back = run(lie_bin, xis)
    for x, t, p, b in zip(xis, T, P, back):
        E = expm(hat6(x))
        assert np.abs(t - E[:3, :4]).max() < 5e-15 * max(1.0, np.abs(E).max())
        assert np.abs(p - np.eye(4)[:3, :4]).max() < 1e-14          # T T^-1
        if np.linalg.norm(x[3:]) < 3.0:
            assert np.abs(b - x).max() < 2e-9 * max(1.0, np.abs(x).max())          # log(exp(xi)): the host log goes through atan2 / tan


def test_se3_exp_is_continuous_across_the_switch(lie_bin):
    u = np.array([0.3, -0.2, 0.1]); d = np.array([0.6, -0.64, 0.48])          # unit direction
    th = 0.5
    xs = [np.concatenate([u, d * (th - 1e-12)]), np.concatenate([u, d * (th + 1e-12)])]
    T, _, _ = run(lie_bin, xs)
    assert np.abs(T[0] - T[1]).max() < 1e-11
