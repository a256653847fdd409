"""
Radiosity solve for a discretised enclosure (reference: emissive_losses/emissive_losses.py:5-83).  A dense
n x n system with n = number of view-factor elements (tens): host work, as in the reference.
"""
import numpy as N

SIGMA = 5.6677e-8       # the Stefan-Boltzmann constant as the reference writes it (emissive_losses.py:29)


def radiosity_RTVF(VF, areas, eps, T=None, inc_radiation=None, q_net=None):
    """
    VF: (n, n) view-factor matrix; areas: (n,) element areas; eps: (n,) emissivities; T: (n,) temperatures with NaN where
    the flux is imposed; inc_radiation: (n,) incident flux densities with NaN where the temperature is imposed; q_net: net
    flux densities removed from the elements.  Returns AA, bb, J, E, T, q, Q as the reference does (system matrix and
    right-hand side, radiosities, black-body emissive power, temperatures, net flux density, net power).

    Same semantics as the reference, including two of its habits: when `inc_radiation` is given only the flux rows of the
    system are filled (:46-51, the temperature rows stay J_i = 0), and T is updated in place for flux rows (:78).
    `q_net` is tested with `is not None` (the reference's `!= None` on an array cannot be evaluated).
    """
    areas = N.asarray(areas, dtype=float)
    n = N.shape(VF)[0]
    if len(eps) != len(areas):
        raise AttributeError
    if T is None and inc_radiation is None:
        raise AttributeError
    system = N.eye(n)
    rhs = N.zeros(n)
    if inc_radiation is not None and T is not None:
        undefined = N.logical_and(N.isnan(T), N.isnan(inc_radiation))
        if undefined.any():
            raise AttributeError('At least one element has no boundary condition for radiosity')
        both = N.logical_and(~N.isnan(T), ~N.isnan(inc_radiation))
        if both.any():
            raise AttributeError('At least one element has two boundary condition definitions for radiosity')
    if inc_radiation is not None:
        flux_rows = ~N.isnan(inc_radiation)
        rhs[flux_rows] += inc_radiation[flux_rows]
        system[flux_rows] += -VF[flux_rows]
    else:
        temp_rows = ~N.isnan(T)
        rhs[temp_rows] += eps * SIGMA * T[temp_rows] ** 4.
        system[temp_rows] += -VF[temp_rows] * (1. - N.vstack(eps[temp_rows]))
    if q_net is not None:
        removed = ~N.isnan(q_net)
        rhs[removed] -= q_net[removed]
    if N.isnan(rhs).any():
        raise AttributeError('Wrong right hand side')
    if N.isnan(system).any():
        raise AttributeError('Wrong system matrix')

    J = N.linalg.solve(system, rhs)

    q = N.zeros(n)
    E = N.zeros(n)
    for i in range(n):
        if ~N.isnan(T[i]):
            E[i] = SIGMA * T[i] ** 4.
            if eps[i] != 1.:
                q[i] = eps[i] / (1. - eps[i]) * (E[i] - J[i])
            else:
                q[i] = E[i] - N.sum(VF[i, :] * J)
        elif ~N.isnan(inc_radiation[i]):
            q[i] = rhs[i]
            T[i] = (1. / SIGMA * (J[i] + (1. - eps[i]) / eps[i] * q[i])) ** 0.25
    E = SIGMA * T ** 4.
    Q = areas * q
    return system, rhs, J, E, T, q, Q
