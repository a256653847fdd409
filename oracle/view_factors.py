"""
TEST INFRASTRUCTURE -- CPU restatement of the reference's emissive_losses package (view-factor statistics, view-factor
allocation of hits, radiosity solve).  Only tests/ may import it.

Pinned by tests/golden/emissive.npz: radiosity() against the reference's own radiosity_RTVF; precision_step() against the
reference's RTVF.test_precision (lines 20-112 of view_factors_3D.py executed from where they lie -- the rest of that file
is Python 2 and does not import).  alloc_two_n() restates Two_N_parameters_cavity_RTVF.alloc_VF (:598-674), which cannot
be run here (Python 2, and it calls source constructors with arguments the reference's sources.py no longer has):
parity unpinned for that function; it is anchored on the text-book view-factor matrices the reference keeps in
emissive_losses_test.py:12-15 and :38-42.
"""
import numpy as N

STEFAN_BOLTZMANN = 5.6677e-8     # emissive_losses.py:29


def radiosity(VF, areas, eps, T=None, inc_radiation=None):
    """emissive_losses.py:5-83 without q_net.  Returns dict(AA, bb, J, E, T, q, Q)."""
    n = VF.shape[0]
    AA = N.zeros((n, n))
    bb = N.zeros(n)
    AA[N.diag_indices(n)] = 1.                                                  # :35
    if inc_radiation is not None:                                               # :46-48
        sel = ~N.isnan(inc_radiation)
        bb[sel] += inc_radiation[sel]
        AA[sel] += -VF[sel]
    else:                                                                       # :49-51
        sel = ~N.isnan(T)
        bb[sel] += eps * STEFAN_BOLTZMANN * T[sel] ** 4.
        AA[sel] += -VF[sel] * (1. - N.vstack(eps[sel]))
    J = N.linalg.solve(AA, bb)                                                  # :62
    T = T.copy()
    q = N.zeros(n)
    for i in range(n):                                                          # :68-78
        if not N.isnan(T[i]):
            Ei = STEFAN_BOLTZMANN * T[i] ** 4.
            q[i] = eps[i] / (1. - eps[i]) * (Ei - J[i]) if eps[i] != 1. else Ei - N.sum(VF[i, :] * J)
        elif not N.isnan(inc_radiation[i]):
            q[i] = bb[i]
            T[i] = (1. / STEFAN_BOLTZMANN * (J[i] + (1. - eps[i]) / eps[i] * q[i])) ** 0.25
    E = STEFAN_BOLTZMANN * T ** 4.                                              # :80
    return dict(AA=AA, bb=bb, J=J, E=E, T=T, q=q, Q=areas * q)


def precision_start(n):
    return dict(VF_esperance=N.zeros((n, n)), Qsum=N.zeros((n, n)), p=N.zeros(n))


def precision_step(state, VF, ray_counts, areas, option, precision, precision_rec=None):
    """
    One call of RTVF.test_precision (view_factors_3D.py:44-112) after `state['p'] += ray_counts` (:208, :545).
    Returns the new state with stdev_VF and progress.
    """
    if precision_rec is None:
        precision_rec = precision                                               # :30-33
    p_tot = state['p'] + ray_counts
    r = N.vstack(ray_counts)                                                    # :51-53
    p = N.vstack(p_tot)
    p_1 = p - r
    Ai = N.ones(VF.shape) * N.vstack(areas)                                     # :56
    with N.errstate(all='ignore'):
        Qsum = state['Qsum'] + r * p_1 / p * (VF - state['VF_esperance']) ** 2.     # :59
        stdev = 3. * N.sqrt(Qsum / (p - 1.)) / N.sqrt(p)                            # :60
        esp = (state['VF_esperance'] * p_1 + VF * r) / p                            # :63
        AiFij = esp * Ai                                                            # :66
        if option == 'absolute':                                                    # :71-76
            stdev_test = stdev <= precision / 2.
            tas = stdev * Ai
            rec_test = (tas + tas.T) <= precision_rec
        else:                                                                       # :79-96
            rel = stdev / esp
            rel[N.isnan(rel)] = 0.
            stdev_test = rel <= precision
            tas = Ai * stdev
            rel_rec = (tas + tas.T) / AiFij
            rel_rec[N.isnan(rel_rec)] = 0.
            rel_rec[N.isinf(rel_rec)] = 0.
            rec_test = N.logical_or(rel_rec <= precision_rec, AiFij < N.vstack(precision_rec * N.amax(AiFij, axis=1)))
        summ_test = N.abs(N.sum(esp, axis=1) - 1.) < precision                      # :98
        progress = N.logical_not(N.logical_and(summ_test, N.logical_and(stdev_test, rec_test)))      # :103
    return dict(VF_esperance=esp, Qsum=Qsum, p=p_tot, stdev_VF=stdev, progress=progress)


def alloc_two_n(hit_surf, hit_abs, hit_pos, apertureRadius, frustaRadii, frustaDepths, el_FRUs, el_CON):
    """
    One row of the pass matrix from the hits of a trace (view_factors_3D.py:598-674).  hit_surf: surface index of every
    hit (0 aperture, 1..len(el_FRUs) the sections, last the cone), hit_abs: absorbed energy, hit_pos: (3, n) positions.
    """
    n_sec = len(el_FRUs)
    row = N.zeros(1 + int(N.sum(el_FRUs)) + int(el_CON))
    heights = N.add.accumulate(N.hstack([0, frustaDepths]))                     # :625
    rads = N.hstack([apertureRadius, frustaRadii])                              # :626
    row[0] = N.sum(hit_abs[hit_surf == 0])                                      # :633
    for j in range(1, n_sec + 1):                                               # :635-660
        sel = hit_surf == j
        e, pos = hit_abs[sel], hit_pos[:, sel]
        hr = N.around(N.sqrt(pos[0] ** 2. + pos[1] ** 2.), decimals=9)
        hh = N.around(pos[2], decimals=9)
        first = 1 + int(N.sum(el_FRUs[:j])) - int(el_FRUs[j - 1])
        for i in range(int(el_FRUs[j - 1])):
            hb = heights[j - 1] + i * (heights[j] - heights[j - 1]) / el_FRUs[j - 1]
            ht = heights[j - 1] + (i + 1) * (heights[j] - heights[j - 1]) / el_FRUs[j - 1]
            r0 = rads[j - 1] + i * (rads[j] - rads[j - 1]) / el_FRUs[j - 1]
            r1 = rads[j - 1] + (i + 1) * (rads[j] - rads[j - 1]) / el_FRUs[j - 1]
            if hb > ht:
                hb, ht = ht, hb
            if r0 > r1:
                r0, r1 = r1, r0
            inside = N.logical_and(N.logical_and(hh >= hb, hh <= ht), N.logical_and(hr >= r0, hr <= r1))
            row[first + i] = N.sum(e[inside])
    sel = hit_surf == n_sec + 1                                                 # :662-672
    e, pos = hit_abs[sel], hit_pos[:, sel]
    cr = N.sqrt(pos[0] ** 2 + pos[1] ** 2)
    for i in range(int(el_CON)):
        r1 = frustaRadii[-1] - i * frustaRadii[-1] / float(el_CON)
        r2 = frustaRadii[-1] - (i + 1) * frustaRadii[-1] / float(el_CON)
        row[1 + int(N.sum(el_FRUs)) + i] = N.sum(e[N.logical_and(cr < r1, cr >= r2)])
    return row
