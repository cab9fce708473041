"""A user objective in the form include/smmhip.h prescribes — an AR(1) simulation with three moments.  The same text
is compiled by hiprtc for the device and by gcc for the oracle (tests only); it is pure arithmetic, so both agree to
the bit."""

AR1_SOURCE = r"""
SMM_USER_OBJECTIVE(const double* theta, int np, const double* mom, const double* w, int nm,
                   const double* udata, int n_udata, double* sim_moments, double* value, int* status)
{
    /* y_t = rho y_{t-1} + sigma e_t with a fixed shock stream; moments: mean(y), mean(y^2), mean(y_t y_{t-1}) */
    const double rho = theta[0], sig = theta[1];
    const int T = (int)udata[0];
    unsigned long long st = 12345ull;
    double y = 0.0, yp = 0.0, s1 = 0.0, s2 = 0.0, s12 = 0.0;
    for (int t = 0; t < T; ++t) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const double u = (double)(st >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        yp = y;
        y = rho * y + sig * u;
        s1 += y; s2 += y * y; s12 += y * yp;
    }
    sim_moments[0] = s1 / T;
    if (nm > 1) sim_moments[1] = s2 / T;
    if (nm > 2) sim_moments[2] = s12 / T;
    if (n_udata > 1 && theta[0] > udata[1]) { *status = -2; *value = -1.0; return; }   /* the model "fails" here */
    double v = 0.0;
    for (int k = 0; k < nm; ++k) { const double d = (sim_moments[k] - mom[k]) / w[k]; v += d * d; }
    *value = v / nm;
    *status = 1;
}
"""


def ar1_numpy(theta, mom, w, udata):
    """independent restatement for the CPU-only test"""
    import numpy as np
    rho, sig = float(theta[0]), float(theta[1])
    T = int(udata[0])
    st = 12345
    y = 0.0; s1 = s2 = s12 = 0.0
    for _ in range(T):
        st = (st * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        u = float(st >> 11) * (1.0 / 9007199254740992.0) - 0.5
        yp = y
        y = rho * y + sig * u
        s1 += y; s2 += y * y; s12 += y * yp
    sm = np.array([s1 / T, s2 / T, s12 / T])[:len(mom)]
    if len(udata) > 1 and rho > udata[1]:
        return sm, -1.0, -2
    d = (sm - np.asarray(mom)) / np.asarray(w)
    return sm, float((d * d).sum() / len(mom)), 1


# The map-reduce form: a panel of AR(1) agents; lane l simulates agents l, l + n_lanes, ...; three sums.
PANEL_SOURCE = r"""
SMM_USER_PARTIAL(const double* theta, int np, const double* udata, int n_udata, int lane, int n_lanes, double* partial)
{
    const double rho = theta[0], sig = theta[1];
    const int T = (int)udata[0], A = (int)udata[1];
    for (int a = lane; a < A; a += n_lanes) {
        unsigned long long st = 12345ull + 7919ull * (unsigned long long)a;
        double y = 0.0, yp = 0.0;
        for (int t = 0; t < T; ++t) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            const double u = (double)(st >> 11) * (1.0 / 9007199254740992.0) - 0.5;
            yp = y;
            y = rho * y + sig * u;
            partial[0] += y; partial[1] += y * y; partial[2] += y * yp;
        }
    }
}

SMM_USER_FINISH(const double* theta, int np, const double* totals, int n_sums, const double* mom, const double* w, int nm,
                const double* udata, int n_udata, double* sim_moments, double* value, int* status)
{
    const double n = udata[0] * udata[1];
    double v = 0.0;
    for (int k = 0; k < nm; ++k) {
        sim_moments[k] = totals[k] / n;
        const double d = (sim_moments[k] - mom[k]) / w[k];
        v += d * d;
    }
    *value = v / nm;
    *status = (n_udata > 2 && theta[0] > udata[2]) ? -2 : 1;
    if (*status < 0) *value = -1.0;
}
"""
