"""
Running estimate of a Monte-Carlo mean from batches of unequal size, with its confidence interval -- the helper the
reference ships as ray_trace_utils/estimator.py:3-56 and that loops of the kind "trace batches until the interval is
narrow enough" are written against (view_factors_3D.py:20-112 has its own copy of the same recurrence).

Written from the statistics, not from that file: with batch means x_k of weights w_k (the rays per batch),

    W = sum w_k,  W2 = sum w_k^2,  mean = sum w_k x_k / W,  S = sum w_k (x_k - mean)^2   (kept by West's update),
    variance = S / (W - W2 / W)   (reliability weights),   effective batches = W^2 / W2,
    half width = n_sigmas * sqrt(variance / effective batches),   relative: divided by the mean.

Same names, arguments and return values as the reference's class and function, so scripts that import it through
`tracer_amd.compat` run as they are; a regression test pins both against a direct evaluation of the formulas above.
"""
import numpy as N


class Estimator(object):
    def __init__(self, n_sigmas=3., relative_CI=True):
        self.n_sigmas = n_sigmas
        self.relative_CI = relative_CI
        self.n = 0.             # W
        self.n2 = 0.            # W2
        self.mean = N.zeros(1)
        self.M2 = N.zeros(1)    # S

    def update(self, values, num_samples):
        x = N.asarray(values, dtype=float)
        w = float(num_samples)
        total = self.n + w
        step = x - self.mean                    # West (1979): mean += w/W' (x - mean); S += w (x - mean_old)(x - mean_new)
        mean = self.mean + step * (w / total)
        self.M2 = self.M2 + w * step * (x - mean) if self.n > 0. else w * step * (x - mean)
        self.mean = mean
        self.n = total
        self.n2 += w * w

    def get_CI(self):
        shape = N.shape(self.mean)
        if self.n == 0.:
            return N.full(shape, N.inf)
        dof = self.n - self.n2 / self.n
        if dof <= 0.:                           # one batch: no spread to speak of yet
            return N.full(shape, N.inf)
        sd = N.sqrt(self.M2 / dof)
        half = self.n_sigmas * sd / N.sqrt(self.n * self.n / self.n2)
        if self.relative_CI:
            with N.errstate(divide='ignore', invalid='ignore'):
                half = half / self.mean
        half = N.array(half, dtype=float, ndmin=1)
        half[N.ravel(sd) == 0.] = 0.
        return half.reshape(shape) if shape else half


def MCRT_to_CI(fun, target_CI, num_samples, n_sigmas=3., *args, **kwargs):
    """call fun(num_rays=num_samples, *args, **kwargs) -> batch mean until the (relative) interval of the running mean is
    below target_CI; returns the Estimator"""
    est = Estimator(n_sigmas)
    while N.any(est.get_CI() > target_CI):
        est.update(fun(num_rays=num_samples, *args, **kwargs), num_samples=num_samples)
    return est
